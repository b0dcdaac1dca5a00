#!/bin/bash
# full GPU test suite, then an interleaved A/B of the default kernel against VARIANT_KEEP_BEHIND_ADDS (1024) at cfg3
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/gpu_sweep.py --workload cfg3 --variants 0,1024 --rounds 5 --tag r01z > gpurun_out/sweep_r01z.log 2>&1; echo "sweep rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/sweep_r01z.log"):
    try: d = json.loads(l)
    except Exception: continue
    print(d["scene"], d["variant"], round(d["median_ms"], 3), round(d["min_ms"], 3))
PY
