#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for w in cfg3 cfg2; do
timeout -k 10 600 python tools/gpu_sweep.py --workload $w --variants 4096,224 --rounds 5 --tag r01zf > gpurun_out/sweep_r01zf_$w.log 2>&1; echo "sweep rc=$?"
python - $w <<'PY'
import json, sys
for l in open("gpurun_out/sweep_r01zf_%s.log" % sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print(sys.argv[1], d["scene"], d["variant"], round(d["median_ms"], 4), round(d["min_ms"], 4))
PY
done
