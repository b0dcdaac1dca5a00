#!/bin/bash
# The hole-density sweep (round 5): cfg 3 with f of the pixels invalidated, scattered (speckle) and in discs (blobs);
# one bench.py line per point -> gpurun_out/<tag>_hole_sweep.jsonl.  usage: tools/gpu_hole_sweep.sh <tag> [workload] [fractions]
set -u
TAG=${1:-r19h}; WL=${2:-cfg3}; FR=${3:-"0 0.001 0.005 0.01 0.02 0.05 0.1 0.2 0.4"}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_hole_sweep.jsonl; : > $OUT
for sc in speckle blobs; do
for f in $FR; do
  timeout -k 10 300 python bench.py --workload $WL --scene $sc --hole-fraction $f --steps 6 --warmup 2 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes 2> gpurun_out/${TAG}_hs.err | tail -1 > gpurun_out/${TAG}_hs.json || { echo "bench failed at $sc $f"; tail -5 gpurun_out/${TAG}_hs.err; exit 1; }
  python - gpurun_out/${TAG}_hs.json $sc $f >> $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(json.dumps({"scene": sys.argv[2], "hole_fraction": float(sys.argv[3]), "ms_per_step": d["ms_per_step"], "kernel_ms": r.get("kernel_ms"), "fuse_ms": r.get("fuse_ms"),
                  "brick_classes": d.get("brick_classes"), "mixed_reasons": d.get("mixed_reasons"), "window_pairs": d.get("window_pairs")}))
PY
  tail -1 $OUT | cut -c1-200
done
done
