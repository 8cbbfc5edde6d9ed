TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $line -d $OUT/pmc$i -o pmc$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --single "$@" > /dev/null 2> $OUT/pmc$i.err
  echo "pmc$i exit $? ($line)"
done <<'PASSES'
TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS
TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ
TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES
TCP_TCP_TA_ADDR_STALL_CYCLES TCP_GATE_EN1
GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES
PASSES
cd $R
python3 scripts/summarize_prof.py $OUT | awk '/reduce_kernel|assemble_kernel/{skip=1} /trace_kernel|== /{skip=0} !skip' | grep "mean/dispatch"
