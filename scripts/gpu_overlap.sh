#!/bin/bash
# do consecutive frames on two streams overlap?  kernel trace of bench --frames-in-flight 2, start/end of the trace_kernel dispatches
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/overlap; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -o ov --output-format csv -- python3 $R/bench.py --config C2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-e2e --frames-in-flight 2 > $OUT/bench.json 2> $OUT/err.log
cd $R; python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/overlap/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "trace_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-8:]:
    print("queue", r.get("Queue_Id"), "stream", r.get("Stream_Id"), "start %.3f ms end %.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
PY
