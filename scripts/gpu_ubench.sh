#!/bin/bash
# SIMD issue cost per instruction class (scripts/ubench/valu_cost.hip) and the exec-mask test -> gpurun_out/ubench_valu_cost.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
[ -x scripts/ubench/valu_cost ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o scripts/ubench/valu_cost scripts/ubench/valu_cost.hip || exit 1
timeout -k 10 300 scripts/ubench/valu_cost > gpurun_out/ubench_valu_cost.txt 2>&1; echo "valu_cost exit $?"; cat gpurun_out/ubench_valu_cost.txt
