#!/bin/bash
# Runs on the GPU box (through gpurun): the randomised differential tests against the oracle on the final build of a round
#   gpurun --timeout 1200 -- 'bash scripts/gpu_fuzz_round.sh [part] [round tag]'
#   part a: wilcoxon + wide gram + wide model; b: generations (+ large); c: sharded, weights, resample, ranking
set -u
OUT=gpurun_out
PART=${1:-abc}
R=${2:-r06}
mkdir -p $OUT
run() { # script, out name, cases, seed, [env]
  timeout -k 10 ${T:-900} python3 tests/fuzz/$1 $OUT/${R}_$2.json $3 $4 > $OUT/fuzz_$2.log 2>&1; tail -1 $OUT/fuzz_$2.log
}
if [[ $PART == *a* ]]; then
run wilcoxon_fuzz.py wilcoxon_fuzz 100 605
run wide_gram_fuzz.py wide_gram_fuzz 40 606
FUZZ_BIG=1 run wide_model_fuzz.py wide_model_fuzz_big 40 607      # (the default's own regime: >= 450 000 rows in every partition)
fi
if [[ $PART == *b* ]]; then
run generation_fuzz.py generation_fuzz 200 608
FUZZ_LARGE=1 run generation_fuzz.py generation_fuzz_large 40 609
fi
if [[ $PART == *c* ]]; then
run sharded_fuzz.py sharded_fuzz 24 610
run weights_fuzz.py weights_fuzz 200 611
run resample_fuzz.py resample_fuzz 150 612
run ranking_fuzz.py ranking_fuzz 150 613
fi
