#!/bin/bash
# PMC passes over the pair-sum kernel alone (scripts/kde_time.py): where the waves' cycles go
#   gpurun -- 'bash scripts/pmc_kde.sh [P 16]'
R=$(pwd); export TMPDIR=/tmp
P="${1:-16}"
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  rm -rf gpurun_out/pmck_$i
  (cd /tmp && KDE_REPS=3 timeout -k 10 200 rocprofv3 --pmc $grp -d $R/gpurun_out/pmck_$i -o p --output-format csv -- python3 $R/scripts/kde_time.py 100000 100000 $P > $R/gpurun_out/pmck_$i.log 2>&1)
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmck_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'k_kde_split' in r['Kernel_Name']:
            tot[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in tot.items():
    print("%-32s n=%d mean %.4g" % (k, len(v), sum(v) / len(v)))
PY
