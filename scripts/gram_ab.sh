#!/bin/bash
# A/B of the 49..96-column statistics kernel (k_gram_dma8) with and without the non-temporal hint: time and FETCH_SIZE
#   gpurun -- 'bash scripts/gram_ab.sh'
export TMPDIR=/tmp ABC_DIAG=1
R=$(pwd)
for v in cached nt; do
  [ $v = nt ] && export ABC_GRAM_DMA8_NT=1
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/gab_f_$v -o x --output-format csv -- python3 $R/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > /dev/null 2>&1
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gab_s_$v -o x --output-format csv -- python3 $R/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-extra) > /dev/null 2>&1
  echo "== $v"
  python3 - <<P
import csv, glob
f = glob.glob("$R/gpurun_out/gab_f_$v/**/*counter_collection.csv", recursive=True)[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'k_gram_dma8' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
print("FETCH_SIZE raw KiB per launch %.0f -> x2 = %.2f GB (algorithmic 7.68 GB)" % (sum(v) / len(v), 2 * 1024 * sum(v) / len(v) / 1e9))
P
  python3 scripts/kstats.py $(find $R/gpurun_out/gab_s_$v -name "*kernel_stats.csv" | head -1) k_gram_dma8
  rm -rf $R/gpurun_out/gab_f_$v $R/gpurun_out/gab_s_$v
done
