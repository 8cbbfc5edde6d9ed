#!/bin/bash
# round 2, GPU call A: microbenchmark + new tests + full GPU suite + default bench
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
./scripts/ubench/exec_mask > gpurun_out/exec_mask.log 2>&1; echo "ubench exit $?"; cat gpurun_out/exec_mask.log
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/test_round2.log 2>&1; echo "round2 tests exit $?"; tail -15 gpurun_out/test_round2.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; tail -c 3000 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
