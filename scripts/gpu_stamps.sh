#!/bin/bash
# diagnostic: phase shares of the trace kernel's wave time (needs `make -C raytrace_clj_amd/csrc stamps`)
set -e
mkdir -p gpurun_out
export RTMI_LIB=$PWD/raytrace_clj_amd/lib/librtmi_stamps.so
for cfg in C2 C3; do
  for accel in bvh flat; do
    [ "$cfg$accel" = "C3flat" ] && continue
    echo "== $cfg $accel" >> gpurun_out/stamps.log
    timeout -k 10 200 python bench.py --config $cfg --accel $accel --single --steps 2 --warmup 1 --no-cpu-baseline >> gpurun_out/stamps.log 2>&1
  done
done
tail -40 gpurun_out/stamps.log
