mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/pytest_gpu.log
for b in 2 3 4 5 6 8; do
  echo "blocks_per_cu $b"
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --blocks-per-cu $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['valu']['frac'])"
done
for c in C2m C3; do echo "config $c"; timeout -k 10 400 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['valu']['frac'], d['config']['segments_per_sample'])"; done
