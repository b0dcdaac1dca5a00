#!/bin/bash
# The geo-referenced scene beside the centred one (cfg 3, speckle), and the view paths.  usage: tools/gpu_geo.sh <tag>
set -u
TAG=${1:-r19q}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_geo.jsonl; : > $OUT
for sc in speckle geo; do
  timeout -k 10 300 python bench.py --workload cfg3 --scene $sc --steps 6 --warmup 2 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes 2> gpurun_out/${TAG}_g.err | tail -1 > gpurun_out/${TAG}_g.json || { tail -5 gpurun_out/${TAG}_g.err; exit 1; }
  python - gpurun_out/${TAG}_g.json $sc >> $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps({"scene": sys.argv[2], "ms_per_step": round(d["ms_per_step"], 3), "kernel_ms": round(d["roofline"]["kernel_ms"], 3), "view_paths": d.get("view_paths"),
                  "brick_classes": d["brick_classes"], "mixed_reasons": d["mixed_reasons"], "window_pairs": d["window_pairs"]}))
PY
  tail -1 $OUT
done
