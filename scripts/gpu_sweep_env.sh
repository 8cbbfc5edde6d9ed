#!/bin/bash
# sweep one environment variable over values: bash scripts/gpu_sweep_env.sh VAR v1 v2 ...   (C2 + C3 ms per step for each)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  bash scripts/gpu_quick.sh "$VAR=$v" || exit 1
done
