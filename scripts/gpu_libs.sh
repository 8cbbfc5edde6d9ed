#!/bin/bash
# diagnostic: time builds of the library (RTMI_LIB) on a few configurations.  usage: [CFGS="C2:bvh C3:bvh"] [EXTRA="..."] gpu_libs.sh lib.so ...
mkdir -p gpurun_out; rm -f gpurun_out/libs.log
CFGS=${CFGS:-"C2:bvh C2:flat C3:bvh CB:bvh FINAL:bvh"}
for lib in "$@"; do
  for cfg in $CFGS; do
    c=${cfg%%:*}; a=${cfg##*:}
    echo "== $lib $c $a" >> gpurun_out/libs.log
    RTMI_LIB=$PWD/raytrace_clj_amd/lib/$lib timeout -k 10 200 python bench.py --config $c --accel $a --single --steps 6 --warmup 1 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined ms/step', d['ms_per_step'], 'Msamples/s', d['value'], '| serial ms/step', d.get('serial', {}).get('ms_per_step'), 'launch ms', d['roofline']['launch_ms'])" >> gpurun_out/libs.log || exit 1
  done
done
cat gpurun_out/libs.log
