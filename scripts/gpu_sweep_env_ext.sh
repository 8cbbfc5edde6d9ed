#!/bin/bash
# like gpu_sweep_env.sh for the section 8 f3 scenes (CB, FINAL)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
VAR=$1; shift
for v in "$@"; do export $VAR=$v; bash scripts/gpu_quick_ext.sh "$VAR=$v" || exit 1; done
