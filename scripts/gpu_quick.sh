#!/bin/bash
# quick perf check: bench C2 + C3 (no extras) for the library given in RTMI_LIB (default: the in-tree one); optional pytest subset
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
TAG=${1:-quick}
for cfg in C2 C3; do
  timeout -k 10 300 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-e2e > gpurun_out/bench_${TAG}_$cfg.json 2> gpurun_out/bench_${TAG}_$cfg.err || { echo "bench $cfg failed"; tail -5 gpurun_out/bench_${TAG}_$cfg.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_${TAG}_$cfg.json").read().strip().splitlines()[-1])
print("$TAG $cfg: %.3f ms/step  %.1f Msamples/s  launch %.3f ms x %d  seg/sample %.4f" % (d["ms_per_step"], d["value"], d["roofline"]["launch_ms"], d["roofline"]["launches_per_step"], d["config"]["segments_per_sample"]))
PY
done
