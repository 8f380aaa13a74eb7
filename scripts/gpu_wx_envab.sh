#!/bin/bash
# Runs on the GPU box: one diagnostic switch of the Wilcoxon reduction A/B-ed on one shape -- kernel statistics of the sweeps, then
# the bench line of the given config, each without and with the switch
#   gpurun -- 'bash scripts/gpu_wx_envab.sh ABC_WX_T768 "1000000 128 16 32" 4'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp ABC_DIAG=1
VAR="$1"; SHAPE="$2"; CFG="$3"
mkdir -p "$OUT"
for v in off on; do
  if [ $v = on ]; then export "$VAR"=1; else unset "$VAR"; fi
  name=envab_${VAR}_$v
  rm -rf "$OUT/prof_$name"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o run --output-format csv -- \
      python3 "$ROOT/scripts/wx_time.py" $SHAPE 5) > "$OUT/$name.log" 2>&1
  echo "== $VAR $v: $(grep ranking $OUT/$name.log)"
  python3 scripts/kstats.py $(find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1) k_wx_sweep k_wx_bounds k_wx_totals
  python3 bench.py --config $CFG --steps 30 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/$name.bench.json" 2> "$OUT/$name.bench.err" &&
    python3 -c "import json,sys; d=json.loads(open('$OUT/$name.bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['roofline']['frac'])"
done
