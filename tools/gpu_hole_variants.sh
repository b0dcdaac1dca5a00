#!/bin/bash
# cfg 3 speckle at low hole densities under the launch shapes the host could choose: default, windows forced (8- and 16-voxel columns)
set -u
TAG=${1:-r19n}; FR=${2:-"0.0003 0.001 0.005 0.01"}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_hole_variants.jsonl; : > $OUT
for f in $FR; do
for v in 0 524288 528384 4096; do
  timeout -k 10 300 python bench.py --workload cfg3 --scene ${SCENE:-speckle} --hole-fraction $f --variant $v --steps 5 --warmup 2 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes 2> gpurun_out/${TAG}_hv.err | tail -1 > gpurun_out/${TAG}_hv.json || { echo "bench failed at $f $v"; tail -5 gpurun_out/${TAG}_hv.err; exit 1; }
  python - gpurun_out/${TAG}_hv.json $f $v >> $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps({"hole_fraction": float(sys.argv[2]), "variant": int(sys.argv[3]), "ms_per_step": round(d["ms_per_step"], 3), "kernel_ms": round(d["roofline"]["kernel_ms"], 3),
                  "mixed": d["brick_classes"]["mixed"], "free_or_no_depth": d["mixed_reasons"]["free_or_no_depth"], "window_pairs": d["window_pairs"]}))
PY
  tail -1 $OUT
done
done
