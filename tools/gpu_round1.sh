#!/bin/bash
# First GPU pass: parity tests, smoke, variant sweep, bench, rocprof kernel trace.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|Socket|Core|Thread" ; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8; } > gpurun_out/host_info.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
timeout 600 python tools/gpu_sweep.py --workload cfg2 --variants 0,1,2,3,4,8,12 --rounds 5 > gpurun_out/sweep_cfg2.log 2>&1
tail -16 gpurun_out/sweep_cfg2.log
timeout 900 python bench.py --steps 3 --warmup 1 --secondary > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; echo "bench rc=$?"
cat gpurun_out/bench_cfg3.json; tail -3 gpurun_out/bench_cfg3.err
