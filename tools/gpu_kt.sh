#!/bin/bash
# kernel trace of the headline bench: per-kernel average durations.  usage: tools/gpu_kt.sh <tag> [bench args]
set -u
TAG=${1:-kt}; shift || true
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ablation --no-coloration --no-end-to-end --no-scenes $* > $O.kt.log 2>&1; echo "kt rc=$?"
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print("%-70s %4s %9.4f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
