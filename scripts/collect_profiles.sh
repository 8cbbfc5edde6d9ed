#!/bin/bash
# after gpu_prof_default.sh + gpu_prof.sh serial_bvh/serial_flat ran on the GPU box: copy the summaries judged into profiles/
set -e
cd "$(dirname "$0")/.."
for t in bvh flat; do
  cp gpurun_out/prof_serial_$t/summary.txt profiles/round1_serial_${t}_C2_summary.txt
  cp gpurun_out/prof_serial_$t/trace/trace_kernel_stats.csv profiles/round1_serial_${t}_C2_kernel_stats.csv
  python scripts/summarize_prof.py gpurun_out/prof_serial_$t --traffic C2/$t/f64 | tail -1 | cut -c1-120
done
cp gpurun_out/prof_default/trace/trace_kernel_stats.csv profiles/round1_default_C2_kernel_stats.csv
tail -1 gpurun_out/prof_default/bench.json > profiles/round1_bench_C2.json
python - <<'PY'
import json
d = json.load(open("profiles/round1_bench_C2.json"))
print("default bench under rocprof:", d["value"], "Msamples/s", d["ms_per_step"], "ms/step; serial", d["serial"]["ms_per_step"], "launch_ms", d["roofline"]["launch_ms"])
PY
head -3 profiles/round1_serial_bvh_C2_kernel_stats.csv | cut -c1-200
