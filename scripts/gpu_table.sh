#!/bin/bash
# the DESIGN.md section 9 table: every bench configuration (ms per step / trace launch ms / Msamples/s), one frame in flight
mkdir -p gpurun_out; rm -f gpurun_out/table.log
run() { # config accel steps extra...
  local cfg=$1 accel=$2 steps=$3; shift 3
  echo "== $cfg $accel $*" >> gpurun_out/table.log
  timeout -k 10 400 python bench.py --config $cfg --accel $accel --no-extras --no-e2e --steps $steps --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'Msamples/s', d['value'], 'launch ms', d['roofline']['launch_ms'], 'x', d['roofline']['launches_per_step'], 'seg/sample', d['config'].get('segments_per_sample'), 'aabb/seg', d.get('aabb_tests_per_segment'), 'prim/seg', d.get('prim_tests_per_segment'))" >> gpurun_out/table.log || exit 1
}
run C2 bvh 10 && run C2 flat 5 && run C2m bvh 10 && run C2m flat 5 && run C3 bvh 5 && run C3 flat 1 && \
run C2 bvh 10 --precision f32 && run C2 flat 5 --precision f32 && run C4 bvh 2 && run C5 bvh 1 && run CB bvh 5 && run CB flat 5 && run FINAL bvh 5 && run FINAL flat 2 && \
run C2 bvh 10 --frames-in-flight 2 && run C3 bvh 5 --frames-in-flight 2
cat gpurun_out/table.log
