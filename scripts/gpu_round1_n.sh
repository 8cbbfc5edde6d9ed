mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu.log
echo "--- torchrun world 1"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --single 2>gpurun_out/tr1.err | cut -c1-400
echo "--- rehearsal world 2 (shared GPU, gloo)"
RTMI_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --single 2>gpurun_out/tr2.err | cut -c1-500
tail -3 gpurun_out/tr2.err
