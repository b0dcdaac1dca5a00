#!/bin/bash
# Kernel edit check in one GPU call: parity + fuzz, then an interleaved A/B of kernel variants on cfg3 (and cfg2).
# Usage: tools/gpu_ab.sh <tag> <variants, comma separated> [workloads]
set -u
TAG=${1:-ab}; VARS=${2:-0,2048}; WL=${3:-cfg3 cfg2}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
[ $rc -eq 0 ] || exit 1
for w in $WL; do
timeout -k 10 600 python tools/gpu_sweep.py --workload $w --variants $VARS --rounds 7 --tag $TAG > gpurun_out/${TAG}_sweep_$w.log 2>&1; echo "sweep rc=$?"
python - $w $TAG <<'PY'
import json, sys
for l in open("gpurun_out/%s_sweep_%s.log" % (sys.argv[2], sys.argv[1])):
    try: d = json.loads(l)
    except Exception: continue
    print(sys.argv[1], d["scene"], d["variant"], "median", round(d["median_ms"], 4), "min", round(d["min_ms"], 4), "main", round(d.get("main_median_ms", 0), 4))
PY
done
