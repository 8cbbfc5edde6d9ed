#!/bin/bash
# ISA of one kernel instantiation -> /tmp/k.s   (default: the FP64 BVH trace kernel)
cd "$(dirname "$0")/.." && make -C raytrace_clj_amd/csrc asm > /dev/null 2>&1
PAT=${1:-_ZN12_GLOBAL__N_112trace_kernelIdLb0ELi4ELb0ELb0ELb1E}
awk -v pat="^$PAT" '$0 ~ pat {f=1} f{print} /s_endpgm/{if(f) exit}' build/rtmi-hip-amdgcn-amd-amdhsa-gfx950.s > /tmp/k.s
wc -l /tmp/k.s
