mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; cat gpurun_out/bench_default.json
for c in C2m C3; do echo "config $c"; timeout -k 10 400 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['valu']['frac'], d['config']['segments_per_sample'])"; done
python -c "import __graft_entry__ as g; g.smoke()"
