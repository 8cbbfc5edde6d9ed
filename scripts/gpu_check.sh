mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
for c in C2 C2m; do echo "config $c accel bvh"; timeout -k 10 400 python bench.py --config $c --accel bvh --single --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
