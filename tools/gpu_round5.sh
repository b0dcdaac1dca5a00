#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/gpu_sweep.py --workload cfg3 --variants 0,160,224,128,96 --rounds 3 --scenes dense --tag r01d > gpurun_out/sweep_cfg3.log 2>&1
tail -6 gpurun_out/sweep_cfg3.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_bit_exact or tiled" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
