#!/bin/bash
# the Cornell box (20 primitives) under both accelerators: is a tree worth anything at this size?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for acc in bvh flat; do
  timeout -k 10 300 python bench.py --config CB --accel $acc --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/bench_cb_$acc.json 2> gpurun_out/bench_cb_$acc.err || { echo "bench CB $acc failed"; tail -5 gpurun_out/bench_cb_$acc.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_cb_$acc.json').read().strip().splitlines()[-1])
print('CB $acc: %.3f ms/step  %.1f Msamples/s' % (d['ms_per_step'], d['value']))"
done
