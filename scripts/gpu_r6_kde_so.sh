#!/bin/bash
# Runs on the GPU box: the pair sums alone (scripts/kde_time.py) under several builds of the library, interleaved twice
#   gpurun --timeout 600 -- 'bash scripts/gpu_r6_kde_so.sh "64 48" default build_ab/lib_x.so'
set -u
export KDE_REPS=${KDE_REPS:-40}
PS="$1"; shift
for P in $PS; do
  for rep in 1 2; do
    for S in "$@"; do
      if [ "$S" = default ]; then unset ABCSMC_HIP_SO; else export ABCSMC_HIP_SO="$PWD/$S"; fi
      echo -n "[$(basename $S)] "; timeout -k 10 120 python3 scripts/kde_time.py ${KDE_K:-100000} ${KDE_KP:-100000} $P 2>&1 | tail -1
    done
  done
done
