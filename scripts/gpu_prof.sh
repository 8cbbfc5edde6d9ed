# usage: bash scripts/gpu_prof.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/  (kernel trace + PMC passes, each its own run)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
(cd $R && python3 -c "import bench; print(bench.kernel_sha())" > $OUT/kernel_sha.txt)  # the sources this profile is taken with
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-e2e "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace exit $?"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $line -d $OUT/pmc$i -o pmc$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-e2e "$@" > /dev/null 2> $OUT/pmc$i.err
  echo "pmc$i exit $? ($line)"
done <<'PASSES'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM
SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32
SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_IOPS SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ
FETCH_SIZE
WRITE_SIZE GRBM_GUI_ACTIVE
PASSES
cd $R
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | grep -v "reduce_kernel\|assemble_kernel" | head -120
