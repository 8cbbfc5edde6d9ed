#!/bin/bash
# diagnostic: sensitivity of the trace kernel to resident workgroups per CU (latency-bound vs throughput-bound)
mkdir -p gpurun_out; rm -f gpurun_out/occ.log
for cfg in C2 C3; do
for b in 1 2 3 4 6 8; do
  echo "== $cfg accel=bvh blocks_per_cu=$b" >> gpurun_out/occ.log
  timeout -k 10 200 python bench.py --config $cfg --accel bvh --single --steps 3 --warmup 1 --no-cpu-baseline --blocks-per-cu $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_ms'])" >> gpurun_out/occ.log || exit 1
done
done
cat gpurun_out/occ.log
