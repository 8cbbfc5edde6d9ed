mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_gpu.log
bash scripts/gpu_prof.sh flat2 --accel flat --single > gpurun_out/prof_flat2.log 2>&1; grep exit gpurun_out/prof_flat2.log | tr '\n' ' '
bash scripts/gpu_prof.sh bvh2 --accel bvh --single > gpurun_out/prof_bvh2.log 2>&1; grep exit gpurun_out/prof_bvh2.log | tr '\n' ' '
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; cat gpurun_out/bench_default.json
