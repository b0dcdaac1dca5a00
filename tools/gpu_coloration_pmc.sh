#!/bin/bash
# L2-request counters of the coloration kernels, one vertex order per pass.  usage: tools/gpu_coloration_pmc.sh <tag>
set -u
TAG=${1:-colpmc}
mkdir -p gpurun_out
export TMPDIR=/tmp
for order in random mesh reordered; do
  timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${TAG}_$order -- python3 tools/gpu_coloration_pmc.py $order > gpurun_out/${TAG}_$order.log 2>&1; echo "$order rc=$?"
done
python3 - $TAG <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = {}
for order in ("random", "mesh", "reordered"):
    line = [l for l in open(f"gpurun_out/{tag}_{order}.log") if l.startswith("{")]
    rec = json.loads(line[-1]) if line else {}
    tot = collections.defaultdict(float)
    for f in glob.glob(f"gpurun_out/{tag}_{order}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "project_color_kernel" in k or "median" in k:  # (the reordered pass: its key / sort kernels are not the gathers and stay out)
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    rec["counters_project_and_median_kernels"] = dict(tot)
    if rec.get("kernel_ms") and tot.get("TCP_TCC_READ_REQ_sum"):
        rec["l2_request_GBps"] = tot["TCP_TCC_READ_REQ_sum"] * 64 / (rec["kernel_ms"] * 1e-3) / 1e9
    out[order] = rec
json.dump(out, open(f"gpurun_out/{tag}_coloration_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
