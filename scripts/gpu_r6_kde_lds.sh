#!/bin/bash
# Runs on the GPU box: the 33..64-parameter pair sums as built (previous tiles staged in LDS, three chunks up to 48 parameters),
# with four chunks everywhere (ABC_KDE_CHUNKS3=0) and with round 5's register-resident kernel (ABC_KDE_LDS=0); the checksums agree
# to the digits the split kernel's error budget leaves (bit for bit where only the staging differs)
#   gpurun --timeout 600 -- 'bash scripts/gpu_r6_kde_lds.sh "64 48 45 33"'
set -u
export ABC_DIAG=1 KDE_REPS=${KDE_REPS:-40}
PS=${1:-"64 61 49 48 46 45 40 33"}
for P in $PS; do
  for E in A=1 ABC_KDE_CHUNKS3=0 ABC_KDE_LDS=0; do
    echo -n "[$E] "; env $E timeout -k 10 120 python3 scripts/kde_time.py ${KDE_K:-100000} ${KDE_KP:-100000} $P 2>&1 | tail -1
  done
done
