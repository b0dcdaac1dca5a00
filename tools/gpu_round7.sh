#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 --secondary > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo "bench rc=$?"
cat gpurun_out/bench_cfg3.json
