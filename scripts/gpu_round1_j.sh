mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu.log
for p in f64 f32; do echo "precision $p"; timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --precision $p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['valu']['frac'])"; done
