#!/bin/bash
# the DESIGN.md section 9 table: every bench configuration, both acceleration structures (ms per step / trace launch ms / Msamples/s)
mkdir -p gpurun_out; rm -f gpurun_out/table.log
run() { # config accel extra...
  local cfg=$1 accel=$2; shift 2
  echo "== $cfg $accel $*" >> gpurun_out/table.log
  timeout -k 10 400 python bench.py --config $cfg --accel $accel --single --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined ms/step', d['ms_per_step'], 'Msamples/s', d['value'], '| serial ms/step', d.get('serial', {}).get('ms_per_step'), 'launch ms', d['roofline']['launch_ms'], 'x', d['roofline']['launches_per_step'], 'seg/sample', d['config'].get('segments_per_sample'))" >> gpurun_out/table.log || exit 1
}
run C2 bvh && run C2 flat && run C2 flat --scan-variant 0 && run C2 flat --scan-variant 2 && run C2m bvh && run C2m flat && run C3 bvh && run C3 flat && \
run C2 bvh --precision f32 && run C2 flat --precision f32 && run C4 bvh && run C5 bvh && run CB bvh && run CB flat && run FINAL bvh && run FINAL flat
cat gpurun_out/table.log
