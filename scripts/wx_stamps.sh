#!/bin/bash
# phase timing of the Wilcoxon kernels (wilcoxon.hip) from in-kernel s_memtime stamps: builds diagnostic copies of the library with
# -DWX_STAMPS=1 (k_wx_ranks), 2 (k_wx_bin, counting sweep), 3 (k_wx_bin, placing sweep) -- never the product build --, runs
# scripts/wx_time.py on each and prints the WX_STAMPS lines
#   gpurun -- 'bash scripts/wx_stamps.sh [N M P A]'
set -e
R=$(pwd)
mkdir -p build_ab && cd abcsmc_amd/csrc
OBJS=$(ls *.o | grep -v '^wilcoxon.o$')
for K in ${WX_KERNELS:-1 2 3}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -DWX_STAMPS=$K -c wilcoxon.hip -o $R/build_ab/wilcoxon_stamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o $R/build_ab/libabcsmc_hip_stamps.so $OBJS $R/build_ab/wilcoxon_stamps.o -ldl -lpthread
  (cd $R && ABCSMC_HIP_SO=$R/build_ab/libabcsmc_hip_stamps.so python3 scripts/wx_time.py "$@" 2>&1 | grep -E "WX_STAMPS|ranking" | tail -2)
done
