#!/bin/bash
# A/B: bash scripts/gpu_ab.sh name1 name2 ...  (name "main" = librtmi.so, else librtmi_<name>.so); prints C2/C3 ms per step for each
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for n in "$@"; do
  if [ "$n" = main ]; then unset RTMI_LIB; else export RTMI_LIB=$R/raytrace_clj_amd/lib/librtmi_$n.so; fi
  bash scripts/gpu_quick.sh $n || exit 1
done
