mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 500 python bench.py --config FINAL --steps 3 --warmup 1 2>gpurun_out/final.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['segments_per_sample'], d['other_accel']['value'], d['cpu_baseline'])"; tail -2 gpurun_out/final.err
