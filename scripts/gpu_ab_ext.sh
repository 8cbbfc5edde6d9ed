#!/bin/bash
# A/B for the mixed-kind kernels: bash scripts/gpu_ab_ext.sh name1 name2 ...  (name "main" = librtmi.so, else librtmi_<name>.so; "name@VAR=v" also exports VAR=v); CB + FINAL ms per step for each
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for spec in "$@"; do
  n=${spec%%@*}; envs=""; [ "$spec" != "$n" ] && envs=${spec#*@}
  if [ "$n" = main ]; then unset RTMI_LIB; else export RTMI_LIB=$R/raytrace_clj_amd/lib/librtmi_$n.so; fi
  ( [ -n "$envs" ] && export $envs; bash scripts/gpu_quick_ext.sh "$spec" ) || exit 1
done
