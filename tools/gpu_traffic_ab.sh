#!/bin/bash
# Fabric traffic of the fusion kernel for two kernel variants (PMC passes on their own, never with a trace domain).
# Usage: tools/gpu_traffic_ab.sh <tag> <variantA> <variantB>
set -u
TAG=${1:-traffic}; VA=${2:-0}; VB=${3:-16384}
mkdir -p gpurun_out
export TMPDIR=/tmp
for V in $VA $VB; do
  O=gpurun_out/${TAG}_v$V
  P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --variant $V"
  timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/fetch -- $P > $O.fetch.log 2>&1; echo "v$V fetch rc=$?"
  timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -- $P > $O.write.log 2>&1; echo "v$V write rc=$?"
done
python3 - "$TAG" "$VA" "$VB" <<'PY'
import csv, glob, sys, collections, json
tag, va, vb = sys.argv[1:4]
out = {}
for v in (va, vb):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{tag}_v{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fuse_tile_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(x) / len(x) for k, x in acc.items()}
    m["read_GB_scale2"] = m.get("FETCH_SIZE", 0) * 1024 * 2 / 1e9
    m["written_GB"] = m.get("WRITE_SIZE", 0) * 1024 / 1e9
    if m.get("TCC_HIT_sum"):
        m["l2_hit"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    out[v] = m
    print(v, {k: round(x, 4) for k, x in m.items()})
json.dump(out, open(f"gpurun_out/{tag}_traffic_ab.json", "w"), indent=1)
PY
