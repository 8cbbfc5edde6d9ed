# usage: bash scripts/gpu_pmc.sh <tag> <passes-file> [bench args...]  -> gpurun_out/pmc_<tag>/ ; one rocprofv3 --pmc run per line of the passes file
TAG=$1; PASSES=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $line -d $OUT/pmc$i -o pmc$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-e2e "$@" > /dev/null 2> $OUT/pmc$i.err
  echo "pmc$i exit $? ($line)"
done < $R/$PASSES
cd $R
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; grep -A40 "trace_kernel" $OUT/summary.txt | grep -v "reduce_kernel\|assemble_kernel" | head -150
