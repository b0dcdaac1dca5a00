#!/bin/bash
# GPU tests of the new pieces + bench; later steps only run when the earlier ones passed
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_cell_to_point.py tests/test_abi.py -m gpu -x -q > gpurun_out/pytest_new.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/pytest_new.log
tail -5 gpurun_out/pytest_new.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ablation --no-coloration > gpurun_out/bench_c2p.json 2> gpurun_out/bench_c2p.err; echo "bench rc=$?"
cat gpurun_out/bench_c2p.json
