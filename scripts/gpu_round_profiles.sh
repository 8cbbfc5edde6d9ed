#!/bin/bash
# the round's profile set on one GPU box: rocprofv3 kernel trace + PMC passes (scripts/gpu_prof.sh) of every bench configuration, then the phase stamps of the
# diagnostic build and the DESIGN.md section 9 table.  usage: bash scripts/gpu_round_profiles.sh [configs...]   (default: C3 C2 FINAL CB C4 C5)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CFGS=${@:-C3 C2 FINAL CB C4 C5}
for c in $CFGS; do
  echo "== prof $c"; date
  bash scripts/gpu_prof.sh ${c}_bvh --config $c > gpurun_out/prof_${c}.log 2>&1 || { echo "prof $c failed"; tail -5 gpurun_out/prof_${c}.log; }
  grep "exit" gpurun_out/prof_${c}.log | tr '\n' ' '; echo
done
