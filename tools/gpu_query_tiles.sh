#!/bin/bash
# pyramid query width A/B: DMI_QUERY_TILES = 2 | 3 | 5 at cfg3 (default shape), dense and sparse
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for q in 5 7 9; do
  DMI_QUERY_TILES=$q timeout -k 10 400 python tools/gpu_sweep.py --workload ${1:-cfg3} --variants 0 --rounds 5 --tag r01q$q > gpurun_out/sweep_q$q.log 2>&1
  python - $q <<'PY'
import json, sys
for l in open("gpurun_out/sweep_q%s.log" % sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print("query", sys.argv[1], d["scene"], round(d["median_ms"], 4), round(d["min_ms"], 4), d["brick_classes"])
PY
done
