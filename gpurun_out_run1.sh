set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench exit $?"
cat gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
