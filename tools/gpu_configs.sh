#!/bin/bash
# One bench line per BASELINE.json config (the single-GPU share of the multi-GPU ones), dense and sparse scenes.
# Usage: tools/gpu_configs.sh <tag>
set -u
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/configs_$TAG.jsonl
: > $OUT
run() {  # name, extra args...
  local name=$1; shift
  echo "== $name $*" >&2
  timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-ablation --no-coloration --no-end-to-end --no-scenes "$@" 2> gpurun_out/configs_$TAG.$name.err | tail -1 > gpurun_out/configs_$TAG.$name.json
  local rc=$?
  if [ -s gpurun_out/configs_$TAG.$name.json ]; then
    python - "$name" gpurun_out/configs_$TAG.$name.json >> $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
d["config_name"] = sys.argv[1]
print(json.dumps(d))
PY
  else
    echo "{\"config_name\": \"$name\", \"failed\": true}" >> $OUT
    return 1
  fi
}
run cfg1_dense  --workload cfg1 --scene dense --cpu-seconds 5 || exit 1
run cfg2_dense  --workload cfg2 --scene dense --cpu-seconds 10 || exit 1
run cfg2_sparse --workload cfg2 --scene sparse --no-cpu-baseline || exit 1
run cfg3_dense  --workload cfg3 --scene dense --no-cpu-baseline || exit 1
run cfg3_sparse --workload cfg3 --scene sparse --no-cpu-baseline || exit 1
run cfg3vga_dense --workload cfg3vga --scene dense --no-cpu-baseline || exit 1
run cfg3_dense_f64grid --workload cfg3 --scene dense --grid-dtype f64 --no-cpu-baseline || exit 1
run cfg4_share_dense --workload 512x128@640x480 --scene dense --no-cpu-baseline || exit 1
run cfg5_share_dense --workload 1024x64@1920x1080 --scene dense --no-cpu-baseline || exit 1
run cfg5_share_sparse --workload 1024x64@1920x1080 --scene sparse --no-cpu-baseline || exit 1
# the same with depth maps as a best-cost threshold leaves them (10 % of the pixels invalid, scattered)
run cfg1_speckle --workload cfg1 --scene speckle --no-cpu-baseline || exit 1
run cfg2_speckle --workload cfg2 --scene speckle --no-cpu-baseline || exit 1
run cfg3_speckle --workload cfg3 --scene speckle --no-cpu-baseline || exit 1
run cfg3vga_speckle --workload cfg3vga --scene speckle --no-cpu-baseline || exit 1
run cfg4_share_speckle --workload 512x128@640x480 --scene speckle --no-cpu-baseline || exit 1
run cfg5_share_speckle --workload 1024x64@1920x1080 --scene speckle --no-cpu-baseline || exit 1
python - $OUT <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d.get("failed"):
        print(d); continue
    print(d["config_name"], round(d["value"], 1), "Gproj/s", round(d["ms_per_step"], 3), "ms", "hbm_frac", round(d["roofline"]["frac"], 4),
          "c2p", round(d["cell_to_point"]["kernel_ms"], 3), "ms", round(d["cell_to_point"]["achieved"]), "GB/s",
          ("cpu %.3f" % d["cpu_baseline"]["value"]) if "cpu_baseline" in d else "")
PY
