# usage: bash scripts/gpu_prof.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace exit $?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc1.err
echo "pmc1 exit $?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc2 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc2.err
echo "pmc2 exit $?"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc3.err
echo "pmc3 exit $?"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc4 -o pmc4 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc4.err
echo "pmc4 exit $?"
cd $R
find $OUT -name "*.csv" | head -30
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
