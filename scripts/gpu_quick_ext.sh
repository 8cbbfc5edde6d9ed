#!/bin/bash
# CB + FINAL (the section 8 f3 scenes) ms per step for the library in RTMI_LIB (default: in-tree)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
TAG=${1:-ext}
for cfg in CB FINAL; do
  timeout -k 10 300 python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/bench_${TAG}_$cfg.json 2> gpurun_out/bench_${TAG}_$cfg.err || { echo "bench $cfg failed"; tail -5 gpurun_out/bench_${TAG}_$cfg.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_${TAG}_$cfg.json').read().strip().splitlines()[-1])
print('$TAG $cfg: %.3f ms/step  %.1f Msamples/s  seg/sample %.4f' % (d['ms_per_step'], d['value'], d['config']['segments_per_sample']))"
done
