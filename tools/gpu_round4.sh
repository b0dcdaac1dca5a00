#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
timeout 900 python tools/gpu_sweep.py --workload cfg3 --variants 0,32,96 --rounds 3 --scenes dense --tag r01c > gpurun_out/sweep_cfg3.log 2>&1
tail -4 gpurun_out/sweep_cfg3.log
