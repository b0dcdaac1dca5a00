#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_coloration.py tests/test_gpu_filter.py tests/test_cell_to_point.py tests/test_abi.py -m gpu -x -q > gpurun_out/pytest_new.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/pytest_new.log
tail -8 gpurun_out/pytest_new.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ablation > gpurun_out/bench_color.json 2> gpurun_out/bench_color.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_color.json"))
print(json.dumps(d.get("coloration")), d["value"], json.dumps(d.get("cell_to_point")))
PY
