mkdir -p gpurun_out
python -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
for v in 0 1 2; do for b in 1 2 4; do
  echo "variant $v blocks_per_cu $b"
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --scan-variant $v --blocks-per-cu $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['valu']['frac'])"
done; done
