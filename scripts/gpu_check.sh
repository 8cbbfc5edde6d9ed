#!/bin/bash
# GPU tests, then the bench configurations (ms per step / trace launch ms / Msamples/s)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -8 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu_libs.sh librtmi.so
# multi-rank control flow rehearsal (2 ranks share the GPU, gather through gloo on the host; never a measurement)
RTMI_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 2>gpurun_out/rehearsal.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rehearsal n_gpus', d['n_gpus'], d['ms_per_step'], d['value'], d['config']['ns'], d.get('serial'))" || { tail -20 gpurun_out/rehearsal.err; exit 1; }
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 1500 gpurun_out/bench_default.json
