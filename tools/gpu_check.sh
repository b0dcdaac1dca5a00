#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for w in cfg1 cfg2 cfg3 1024x64@1920x1080; do
  timeout -k 10 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-ablation --no-coloration 2> gpurun_out/bench_$w.err | tail -1 > gpurun_out/bench_$w.json
  python - gpurun_out/bench_$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], round(d["value"], 1), "Gproj/s", round(d["ms_per_step"], 4), "ms/step", "kernel_ms", round(d["roofline"]["kernel_ms"], 4))
PY
done
