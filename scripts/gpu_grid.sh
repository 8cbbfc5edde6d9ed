#!/bin/bash
# entry grid experiments: C2 / C3 ms per step and node visits per segment for specs "cells_per_side:kmax:suspend_lanes" (cells 0 = off, 1 = default)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
CFGS=${CFGS:-"C2 C3"}
for spec in "$@"; do
  IFS=: read g k sl <<< "$spec"
  for cfg in $CFGS; do
    RTMI_GRID=$g RTMI_GRID_KMAX=$k RTMI_SUSPEND_LANES=${sl:-8} timeout -k 10 300 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-e2e 2> gpurun_out/grid.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $g kmax $k suspend ${sl:-8} $cfg: %.3f ms/step launch %.3f  visits/seg %.2f exact/seg %.2f' % (d['ms_per_step'], d['roofline']['launch_ms'], d['aabb_tests_per_segment']/2, d['prim_tests_per_segment']))" || { tail -5 gpurun_out/grid.err; exit 1; }
  done
done
