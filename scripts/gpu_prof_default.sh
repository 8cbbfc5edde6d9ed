#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT bench command (pipelined + serial + flat steps + CPU baseline) -> gpurun_out/prof_default/
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_default
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py > $OUT/bench.json 2> $OUT/trace.err
echo "trace exit $?"
cd $R
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; head -8 $OUT/summary.txt
