#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_filter.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
timeout 900 python tools/gpu_sweep.py --workload cfg3 --variants 0,256,32 --rounds 3 --scenes dense,sparse --tag r01e > gpurun_out/sweep_cfg3.log 2>&1
tail -8 gpurun_out/sweep_cfg3.log
