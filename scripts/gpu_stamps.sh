#!/bin/bash
# diagnostic: per-phase wave ticks and lane-ticks of the trace kernel (needs `make -C raytrace_clj_amd/csrc stamps`)
# usage: gpu_stamps.sh <lib> "<config> <accel>"...   -> gpurun_out/stamps.log
mkdir -p gpurun_out
lib=$1; shift
export RTMI_LIB=$PWD/raytrace_clj_amd/lib/$lib
for ca in "$@"; do
  set -- $ca
  echo "== $lib $1 $2" >> gpurun_out/stamps.log
  timeout -k 10 200 python bench.py --config $1 --accel $2 --single --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "stamps\]\|phases\]" | tail -24 >> gpurun_out/stamps.log
done
cat gpurun_out/stamps.log
