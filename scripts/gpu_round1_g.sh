mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/pytest_gpu.log
for lib in librtmi.so librtmi_w3.so librtmi_w2.so; do for b in 2 3 4; do
  echo "lib $lib blocks_per_cu $b"
  RTMI_LIB=$PWD/raytrace_clj_amd/lib/$lib timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --blocks-per-cu $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['valu']['frac'])"
done; done
