#!/bin/bash
# GPU tests, then the bench configurations (ms per step / trace launch ms / Msamples/s)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -8 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu_libs.sh librtmi.so
