#!/bin/bash
# What a kind of (brick, view) pair costs, from counters of the TUNING library (DMI_TUNING=1 build): the headline launch as it
# is, without its window pairs, and with its window pairs alone (results of the last two are wrong: timing only).
# usage: tools/gpu_pair_cost.sh <tag>
set -u
TAG=${1:-r19c}
mkdir -p gpurun_out
export TMPDIR=/tmp DMI_TUNING=1
O=gpurun_out/$TAG
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes"
for mode in all nowin onlywin; do
  if [ $mode = all ]; then unset DMI_DEBUG_PAIRS; else export DMI_DEBUG_PAIRS=$mode; fi
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/$mode -- $P > $O.$mode.log 2>&1; echo "$mode rc=$?"
  grep -o '"ms_per_step": [0-9.]*' $O.$mode.log | head -1
done
python3 - $O <<'PY'
import csv, glob, sys, collections, json
out = {}
for mode in ("all", "nowin", "onlywin"):
    f = glob.glob(sys.argv[1] + "/" + mode + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "fuse_tile_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[mode] = {k: sum(v) / len(v) for k, v in acc.items()}
    print(mode, {k: "%.4g" % v for k, v in out[mode].items()})
json.dump(out, open(sys.argv[1] + "_pair_cost.json", "w"), indent=1)
PY
