#!/bin/bash
# the round-end sequence on one box: full GPU suite, smoke, default bench
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/test_full.log 2>&1; echo "full gpu suite exit $?"; tail -6 gpurun_out/test_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["config"]["workload"][:60], d["value"], "Msamples/s", d["ms_per_step"], "ms; roofline frac", d["roofline"]["frac"], "hbm", d["roofline"]["hbm"]["frac"], "fetch", d["roofline"]["on_chip_fetch"]["frac"])
for k in ("pipelined","c2","c4","cpu_baseline"): print(k, d[k]["value"], d[k].get("ms_per_step"))
PY
