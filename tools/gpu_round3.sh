#!/bin/bash
# GPU pass: parity of the tiled kernel, smoke, tile-shape sweep, bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
timeout 600 python tools/gpu_sweep.py --workload cfg2 --variants 0,32,64,96,16 --rounds 5 --tag r01b > gpurun_out/sweep_cfg2.log 2>&1
tail -12 gpurun_out/sweep_cfg2.log
timeout 900 python bench.py --steps 3 --warmup 1 --secondary > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo "bench rc=$?"
cat gpurun_out/bench_cfg3.json; tail -3 gpurun_out/bench_cfg3.err
