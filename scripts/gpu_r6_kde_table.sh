#!/bin/bash
# Runs on the GPU box: the pair sums by parameter count at 1e10 pairs (sustained clock: 60 repetitions), as built and with round 5's
# register-resident kernel at 33..64 parameters; then the kernel trace of the 64-parameter case
#   gpurun --timeout 900 -- 'bash scripts/gpu_r6_kde_table.sh'
set -u
export ABC_DIAG=1 KDE_REPS=60
OUT=gpurun_out
for P in 16 32 33 40 45 48 49 61 64; do
  timeout -k 10 120 python3 scripts/kde_time.py 100000 100000 $P 2>&1 | tail -1
done > $OUT/r06_kde_by_parameters.txt
echo "-- round 5's kernel (ABC_KDE_LDS=0: previous tiles in registers, one wave per SIMD, four chunks)" >> $OUT/r06_kde_by_parameters.txt
for P in 33 48 64; do
  ABC_KDE_LDS=0 timeout -k 10 120 python3 scripts/kde_time.py 100000 100000 $P 2>&1 | tail -1
done >> $OUT/r06_kde_by_parameters.txt
cat $OUT/r06_kde_by_parameters.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KDE_REPS=20 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_kde64 -o kde64 --output-format csv -- python3 $R/scripts/kde_time.py 100000 100000 64 > $R/$OUT/prof_kde64.log 2>&1
CSV=$(find $R/$OUT/prof_kde64 -name "*kernel_stats.csv" | head -1)
python3 $R/scripts/kstats.py "$CSV" | head -12 > $R/$OUT/r06_kde64_kernel_stats.txt
cat $R/$OUT/r06_kde64_kernel_stats.txt
